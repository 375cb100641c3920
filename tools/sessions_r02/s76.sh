cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s76; mkdir -p $O
SRW_TIMING=1 timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "degrades" -s > $O/fb.txt 2>&1 < /dev/null; grep -E "passed|failed|per-edge tables:" $O/fb.txt | cut -c1-250 | tail -6
