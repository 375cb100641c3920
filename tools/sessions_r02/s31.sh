cd $GRAFT_REPO_ROOT
O=gpurun_out/s31; mkdir -p $O
for v in "" _shr2 _shr8; do
  SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw$v.so timeout 300 python tools/cluster_timing1.py 24 1,8 2>&1 < /dev/null | grep world | tee -a $O/t.txt
done
