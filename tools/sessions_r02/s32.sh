cd $GRAFT_REPO_ROOT
O=gpurun_out/s32; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/pytest.txt 2>&1 < /dev/null; tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 < /dev/null; tail -2 $O/smoke.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --scale 24 --shard vertex > $O/bench_vertex_cluster.txt 2>&1 < /dev/null; tail -1 $O/bench_vertex_cluster.txt | cut -c1-1500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 8 --warmup 2 --scale 24 --shard vertex --shard-driver rccl > $O/bench_vertex_rccl.txt 2>&1 < /dev/null; tail -1 $O/bench_vertex_rccl.txt | cut -c1-1500
