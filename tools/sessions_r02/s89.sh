cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s89; mkdir -p $O
for r in 1 2; do
for v in old new; do
  if [ $v = old ]; then export SRW_LIB=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_old.so; else unset SRW_LIB; fi
  timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3_$v$r.txt 2>&1 < /dev/null; echo "C3 $v: $(grep -E '^iter [12]' $O/c3_$v$r.txt | cut -c1-60 | tr '\n' ' ')"
done
done
