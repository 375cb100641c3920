cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s74; mkdir -p $O
SRW_EB_CHUNKS=512 timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -k "biased or directed or binned or edge_table or rmat_vs_oracle" > $O/parity_512.txt 2>&1 < /dev/null; echo "parity chunks 512: $(grep -E 'passed|failed' $O/parity_512.txt | tail -1)"
for c in 512 256; do SRW_EB_CHUNKS=$c SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3_$c.txt 2>&1 < /dev/null; echo "chunks $c:"; grep -E "^iter [12]|edge tables\]" $O/c3_$c.txt | cut -c1-200; done
