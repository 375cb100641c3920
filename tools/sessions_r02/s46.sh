cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s46; mkdir -p $O
for hb in 40 64; do
  SRW_TIMING=1 SRW_HUB_BUDGET_GB=$hb timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_hub_$hb.txt 2>&1 < /dev/null; echo "C5 hub $hb"; grep -E "^iter 1|edge tables\]" $O/c5_hub_$hb.txt | cut -c1-260
done
