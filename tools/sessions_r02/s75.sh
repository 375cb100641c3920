cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s75
mkdir -p $O
summ() { f=$(find "$1" -name "*$2*.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then python $R/tools/prof_summary.py $3 "$f" $4; else echo "no $2 csv under $1"; fi; }
G="python $R/tools/one_walk.py 24w 0.25 4 reference 1"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/eb_$n -- $G > $O/eb_$n.txt 2>&1 < /dev/null
  summ $O/eb_$n counter_collection counters k_eb_build >> $O/eb_counters.txt
done
cat $O/eb_counters.txt
grep -E "^iter|edge tables\]" $O/eb_FETCH_SIZE.txt | cut -c1-200
find $O -name '*.csv' -delete
