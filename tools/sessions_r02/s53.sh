cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s53; mkdir -p $O
timeout 900 python -m pytest tests/test_properties.py -q -m gpu -x > $O/props.txt 2>&1 < /dev/null; tail -5 $O/props.txt | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --scale 22 --steps 3 --warmup 1 --configs 0 --end-to-end 0 --cpu-baseline 0 > $O/torchrun1.txt 2> $O/torchrun1.err < /dev/null; tail -1 $O/torchrun1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('torchrun world1: value', d['value'], 'vertex_sharded', json.dumps(d.get('vertex_sharded'))[:400])"
timeout 900 python bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 > $O/headline.txt 2> $O/headline.err < /dev/null; tail -1 $O/headline.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
