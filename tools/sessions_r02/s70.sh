cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s70; mkdir -p $O
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep -E "^iter|edge tables\]|edge hash vs" $O/c5.txt | cut -c1-250
SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3.txt 2>&1 < /dev/null; grep -E "^iter|edge tables\]|edge hash vs" $O/c3.txt | cut -c1-250
