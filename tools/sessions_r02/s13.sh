set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s13; mkdir -p $O
SRW_TIMING=1 SRW_SHARD_PROFILE=1 timeout 300 python tools/cluster_timing.py 24 1 > $O/cluster_profile.txt 2>&1 < /dev/null; grep -E "cluster world|profile|overflow|replicated" $O/cluster_profile.txt | tail -6
SRW_TIMING=1 timeout 400 python tools/cluster_timing.py 24 2,8 > $O/cluster_2_8.txt 2>&1 < /dev/null; grep -E "cluster world|overflow" $O/cluster_2_8.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 4 --warmup 1 --scale 24 --configs 0 --end-to-end 0 --cpu-baseline 0 > $O/bench_torchrun.json 2> $O/bench_torchrun.err < /dev/null; grep -o '"vertex_sharded".*' $O/bench_torchrun.json | head -c 400; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 4 --warmup 1 --scale 24 --configs 0 --end-to-end 0 --cpu-baseline 0 --shard vertex --shard-driver rccl > $O/bench_rccl.json 2> $O/bench_rccl.err < /dev/null; tail -2 $O/bench_rccl.err; grep -o '"vertex_sharded".*' $O/bench_rccl.json | head -c 400; echo
