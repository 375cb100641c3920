set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cluster or sharded" > gpurun_out/s8/pytest.txt 2>&1; tail -30 gpurun_out/s8/pytest.txt
timeout 600 python tools/cluster_timing.py 24 1,2,8 > gpurun_out/s8/cluster24.txt 2>&1; cat gpurun_out/s8/cluster24.txt
timeout 600 python tools/cluster_timing.py 20 1,2 0.25 4 > gpurun_out/s8/cluster20q.txt 2>&1; cat gpurun_out/s8/cluster20q.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 1 --scale 24 --configs 0 --end-to-end 0 --cpu-baseline 0 > gpurun_out/s8/bench_torchrun.json 2> gpurun_out/s8/bench_torchrun.err; tail -5 gpurun_out/s8/bench_torchrun.err; cat gpurun_out/s8/bench_torchrun.json
