set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
for s in a b c d; do python tools/torch_after_lib.py $s >> gpurun_out/s6/torch_after.txt 2>&1; done; cat gpurun_out/s6/torch_after.txt
env | grep -i -E "hip|rocr|hsa|cuda|visible" > gpurun_out/s6/env.txt; cat gpurun_out/s6/env.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sharded_kernels" > gpurun_out/s6/alone.txt 2>&1; tail -5 gpurun_out/s6/alone.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sample_kat or sharded_kernels" > gpurun_out/s6/kat.txt 2>&1; tail -5 gpurun_out/s6/kat.txt
