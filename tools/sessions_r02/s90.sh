cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s90; mkdir -p $O
SRW_EB_DROP_EHASH=1 timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_properties.py tests/test_sparse_ids.py -q -m gpu > $O/parity_drop.txt 2>&1 < /dev/null; echo "parity, hash dropped (filters on): $(grep -E 'passed|failed' $O/parity_drop.txt | tail -1)"; grep -E "^FAILED" $O/parity_drop.txt | head
timeout 1800 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; echo "default: $(grep -E 'passed|failed' $O/pytest.txt | tail -1)"; grep -E "^FAILED" $O/pytest.txt | head
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep -E "^iter|row filters|edge hash vs" $O/c5.txt | cut -c1-220
for v in old new; do
  if [ $v = old ]; then export SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_old.so; else unset SRW_LIB; fi
  SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3_$v.txt 2>&1 < /dev/null; echo "C3 $v: $(grep -E '^iter [12]|row filters' $O/c3_$v.txt | cut -c1-60 | tr '\n' ' ')"
done
