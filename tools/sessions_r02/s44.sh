cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s44; mkdir -p $O
for hb in 4 40 80; do
  SRW_HUB_BUDGET_GB=$hb SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_timing.so timeout 900 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/hub_$hb.txt 2>&1 < /dev/null; echo "hub budget $hb"; grep -E "^iter 1|lean" $O/hub_$hb.txt | tail -2 | cut -c1-600
done
