cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s55; mkdir -p $O
timeout 900 python -m pytest tests/test_sparse_ids.py tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -k "sparse or cluster or shard" > $O/sparse.txt 2>&1 < /dev/null; tail -30 $O/sparse.txt | cut -c1-300
for i in 1 2 3; do timeout 600 python bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['roofline']['kernel_ms_avg'])"; done
