cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s77; mkdir -p $O
for c in default 128 512 100; do
  if [ $c = default ]; then unset SRW_EB_CHUNKS; else export SRW_EB_CHUNKS=$c; fi
  timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -k "biased or directed or binned or edge_table or rmat_vs_oracle" > $O/parity_$c.txt 2>&1 < /dev/null; echo "parity chunks $c: $(grep -E 'passed|failed' $O/parity_$c.txt | tail -1)"
done
unset SRW_EB_CHUNKS
SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3.txt 2>&1 < /dev/null; grep -E "^iter [12]|edge tables\]" $O/c3.txt | cut -c1-200
SRW_EB_CHUNKS=512 SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3_512.txt 2>&1 < /dev/null; grep -E "^iter [12]|edge tables\]" $O/c3_512.txt | cut -c1-200
