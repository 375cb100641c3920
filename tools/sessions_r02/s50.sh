cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s50; mkdir -p $O
for c in 32 16; do
  SRW_TIMING=1 SRW_EB_CHUNKS=$c timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_chunks_$c.txt 2>&1 < /dev/null; echo "C5 chunks $c"; grep -E "^iter 1|edge tables\]" $O/c5_chunks_$c.txt | cut -c1-300
done
