cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s79; mkdir -p $O
SRW_TIMING=1 timeout 1500 python tests/big_c3_check.py 24 > $O/c3_full.txt 2>&1 < /dev/null; grep -E "oracle graph|IDENTICAL|MISMATCH|parity|edge hash vs|edge tables\]" $O/c3_full.txt | cut -c1-330
