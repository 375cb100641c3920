cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s49; mkdir -p $O
for v in rpl1 rpl4 lw4 lw6; do
  SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_$v.so timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/$v.txt 2>&1 < /dev/null; echo "$v"; grep -E "^iter [12]" $O/$v.txt | cut -c1-120
done
