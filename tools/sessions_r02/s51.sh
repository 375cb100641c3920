cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s51; mkdir -p $O
for c in 32 16 8; do
  SRW_EB_CHUNKS=$c timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -k "biased or directed or binned or edge_table or rmat_vs_oracle" > $O/parity_chunks_$c.txt 2>&1 < /dev/null; echo "parity chunks $c: $(tail -1 $O/parity_chunks_$c.txt)"
done
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_default.txt 2>&1 < /dev/null; grep -E "^iter 1|edge tables\]" $O/c5_default.txt | cut -c1-300
SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3_default.txt 2>&1 < /dev/null; grep -E "^iter 1|edge tables\]" $O/c3_default.txt | cut -c1-300
