cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s56; mkdir -p $O
smi() { rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Temperature|Power" | sed 's/  */ /g' | tr '\n' ';' | cut -c1-700; echo; }
echo "idle: $(smi)" | tee $O/smi.txt
for i in 1 2 3 4 5; do
  ( sleep 9; echo "during run $i: $(smi)" >> $O/smi.txt ) &
  timeout 600 python bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 --steps 40 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['roofline']['kernel_ms_avg'])" | tee -a $O/smi.txt
  wait
done
sleep 30; echo "after 30 s idle: $(smi)" >> $O/smi.txt
timeout 600 python bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline after idle', d['value'], d['roofline']['kernel_ms_avg'])" | tee -a $O/smi.txt
cat $O/smi.txt
