set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s17; mkdir -p $O
L=$GRAFT_REPO_ROOT/stellar-random-walk_amd
timeout 1800 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -k "scale or rmat or edge_tables or binned or giant or karate or fuzz or weighted or directed or golden" > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED|Error" $O/pytest.txt | head
for v in "" _lw5 _lw6; do
  SRW_LIB=$L/libstellar_rw$v.so timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3$v.txt 2>&1 < /dev/null; echo "variant '$v'"; grep "^iter 1" $O/c3$v.txt
done
SRW_NO_LEAN_KERNEL=1 timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3_nolean.txt 2>&1 < /dev/null; echo "no lean"; grep "^iter 1" $O/c3_nolean.txt
timeout 300 python tools/one_walk.py 20 0.25 4 reference 3 > $O/c20.txt 2>&1 < /dev/null; grep "^iter" $O/c20.txt
SRW_NO_LEAN_KERNEL=1 timeout 300 python tools/one_walk.py 20 0.25 4 reference 3 > $O/c20_nolean.txt 2>&1 < /dev/null; grep "^iter" $O/c20_nolean.txt
