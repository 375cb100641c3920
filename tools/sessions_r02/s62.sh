cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s62; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  SRW_TIMING=1 timeout 600 python bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 --steps 6 --warmup 1 > $O/run$i.txt 2> $O/run$i.err < /dev/null
  echo "run $i: $(grep -o 'bytes at 0x[0-9a-f]*' $O/run$i.err | head -1) $(tail -1 $O/run$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('kernel_ms', round(d['roofline']['kernel_ms_avg'],2))")"
done | tee $O/summary.txt
