cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s43; mkdir -p $O
SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_timing.so timeout 900 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/lean_timing.txt 2>&1 < /dev/null; grep -E "^iter|lean|phase\] wave" $O/lean_timing.txt | cut -c1-600
