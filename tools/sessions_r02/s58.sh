cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s58; mkdir -p $O
timeout 1200 python tests/big_c4_check.py 26 8 16 > $O/c4_26_w8.txt 2>&1 < /dev/null; tail -8 $O/c4_26_w8.txt | cut -c1-300
timeout 1200 python tests/big_c4_check.py 27 4 16 > $O/c4_27_w4.txt 2>&1 < /dev/null; tail -8 $O/c4_27_w4.txt | cut -c1-300
