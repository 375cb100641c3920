cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s93; mkdir -p $O
SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_r4.so timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_r4.txt 2>&1 < /dev/null; echo "C5, 4 candidates per lane and round:"; grep -E "^iter" $O/c5_r4.txt | cut -c1-100
