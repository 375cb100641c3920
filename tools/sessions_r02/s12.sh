set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cluster or sharded" > $O/pytest.txt 2>&1 < /dev/null; tail -12 $O/pytest.txt
SRW_SHARD_PROFILE=1 timeout 300 python tools/cluster_timing.py 24 1 > $O/cluster_profile.txt 2>&1 < /dev/null; grep -E "cluster world|profile" $O/cluster_profile.txt | tail -4
timeout 300 python tools/cluster_timing.py 24 2,8 > $O/cluster_2_8.txt 2>&1 < /dev/null; grep "cluster world" $O/cluster_2_8.txt
L=$GRAFT_REPO_ROOT/stellar-random-walk_amd
for v in "" _r2 _r1; do
  SRW_LIB=$L/libstellar_rw$v.so timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3$v.txt 2>&1 < /dev/null; echo "variant '$v'"; grep "^iter 1" $O/c3$v.txt
done
for v in "" _r2; do
  SRW_LIB=$L/libstellar_rw$v.so timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5$v.txt 2>&1 < /dev/null; echo "variant '$v'"; grep "^iter 1" $O/c5$v.txt
done
