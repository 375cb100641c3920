cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s60
mkdir -p $O
summ() { f=$(find "$1" -name "*$2*.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then python $R/tools/prof_summary.py $3 "$f" $4; else echo "no $2 csv under $1"; fi; }
G="python $R/tools/one_walk.py 24w 0.25 4 reference 2"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3r_trace -- $G > $O/c3r_trace.txt 2>&1 < /dev/null; grep "^iter" $O/c3r_trace.txt | cut -c1-160
summ $O/c3r_trace kernel_trace stats > $O/c3r_kernel_stats.txt; head -8 $O/c3r_kernel_stats.txt
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/c3r_$n -- $G > $O/c3r_$n.txt 2>&1 < /dev/null
  summ $O/c3r_$n counter_collection counters k_walk_tables >> $O/c3r_counters.txt
done
cat $O/c3r_counters.txt
B="python $R/bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 --steps 3 --warmup 1"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/fo_$n -- $B > $O/fo_$n.txt 2>&1 < /dev/null
  summ $O/fo_$n counter_collection counters k_walk_first_order >> $O/fo_counters.txt
done
cat $O/fo_counters.txt
find $O -name '*.csv' -size +8M -delete
