set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s10
mkdir -p $O
summ() { # dir mode [filter] -> summary to stdout; never blocks on a missing file
  f=$(find "$1" -name "*$2*.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/prof_summary.py $3 "$f" $4; else echo "no $2 csv under $1"; ls -R "$1" | head -20; fi
}
# 1. sharded super-step split (debug timers) + cluster leg of bench under torchrun
SRW_SHARD_PROFILE=1 timeout 300 python $R/tools/cluster_timing.py 24 1 > $O/cluster_profile.txt 2>&1 < /dev/null; cat $O/cluster_profile.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --steps 4 --warmup 1 --scale 24 --configs 0 --end-to-end 0 --cpu-baseline 0 > $O/bench_torchrun.json 2> $O/bench_torchrun.err < /dev/null; tail -3 $O/bench_torchrun.err; python -c "import json;j=json.load(open('$O/bench_torchrun.json'));print(j['value'], j['vertex_sharded'])"
# 2. headline bench: kernel trace + stats
B="python $R/bench.py --cpu-baseline 0 --configs 0 --end-to-end 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl_trace -- $B --steps 10 --warmup 2 > $O/hl_trace.txt 2>&1 < /dev/null; tail -1 $O/hl_trace.txt | head -c 600; echo
summ $O/hl_trace kernel_trace stats > $O/hl_kernel_stats.txt; cat $O/hl_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/hl_$n -- $B --steps 3 --warmup 1 > $O/hl_$n.txt 2>&1 < /dev/null
  summ $O/hl_$n counter_collection counters k_walk_first_order >> $O/hl_counters.txt
done
cat $O/hl_counters.txt
