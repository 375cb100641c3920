cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s68; mkdir -p $O
SRW_ONE_WALK_KW='{"edge_hash": false}' SRW_EB_CHUNKS=64 SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_64_noehash.txt 2>&1 < /dev/null; echo "64 chunks, no edge hash:"; grep -E "^iter|edge tables\]" $O/c5_64_noehash.txt | cut -c1-250
SRW_ONE_WALK_KW='{"edge_hash": false}' SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_def_noehash.txt 2>&1 < /dev/null; echo "default chunks, no edge hash:"; grep -E "^iter|edge tables\]" $O/c5_def_noehash.txt | cut -c1-250
