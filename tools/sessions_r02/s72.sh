cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s72; mkdir -p $O
free -g | head -2
SRW_TIMING=1 timeout 1500 python tests/big_c5_check.py 26 27 > $O/c5_full.txt 2>&1 < /dev/null; grep -E "oracle graph|IDENTICAL|MISMATCH|parity|edge hash vs|edge tables\]|Error|error" $O/c5_full.txt | cut -c1-400
