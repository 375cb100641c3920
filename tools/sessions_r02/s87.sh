cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s87; mkdir -p $O
SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_timing.so SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_timing.txt 2>&1 < /dev/null; grep -E "^iter|\[lean\]|chunks|edge hash vs" $O/c5_timing.txt | cut -c1-400
