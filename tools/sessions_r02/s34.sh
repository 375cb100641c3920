cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s34; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 8 --warmup 2 --scale 24 --shard vertex --shard-driver rccl > $O/rccl.txt 2>&1 < /dev/null; echo rc=$?; grep -E "Memory access|\"value\"" $O/rccl.txt | cut -c1-2000
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sharded or cluster" 2>&1 | tail -2
rm -f gpucore.* core.*
