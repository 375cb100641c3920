cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s84; mkdir -p $O
ldd stellar-random-walk_amd/libstellar_rw.so | grep -E "libz|not found"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gz or directory or cli or load" > $O/t.txt 2>&1 < /dev/null; tail -8 $O/t.txt | cut -c1-300
