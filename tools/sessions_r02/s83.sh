cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s83; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "directory or non_ascii or tokenizer or load or cli" > $O/t.txt 2>&1 < /dev/null; tail -8 $O/t.txt | cut -c1-300
