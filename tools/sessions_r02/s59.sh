cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s59; mkdir -p $O
timeout 900 python -m pytest tests/ -q -m gpu -k "cluster or shard" > $O/cl.txt 2>&1 < /dev/null; tail -5 $O/cl.txt | cut -c1-300
timeout 300 bash tests/cli_c2.sh > $O/cli.txt 2>&1 < /dev/null; tail -3 $O/cli.txt | cut -c1-200
timeout 1200 python tests/big_c4_check.py 26 8 16 > $O/c4_26_w8.txt 2>&1 < /dev/null; tail -6 $O/c4_26_w8.txt | cut -c1-300
timeout 1200 python tests/big_c4_check.py 27 4 16 > $O/c4_27_w4.txt 2>&1 < /dev/null; tail -8 $O/c4_27_w4.txt | cut -c1-300
