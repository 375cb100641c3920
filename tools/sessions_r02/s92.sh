cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s92
mkdir -p $O
summ() { f=$(find "$1" -name "*$2*.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then python $R/tools/prof_summary.py $3 "$f" $4; else echo "no $2 csv under $1"; fi; }
G="python $R/tools/one_walk.py 26d 4 0.5 reference 2 27"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "TCC_EA0_RDREQ_sum TCC_HIT_sum FETCH_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/c5r_$n -- $G > $O/c5r_$n.txt 2>&1 < /dev/null
  summ $O/c5r_$n counter_collection counters k_walk_tables >> $O/c5r_counters.txt
done
cat $O/c5r_counters.txt; grep "^iter" $O/c5r_TCC_EA0_RDREQ_sum.txt | cut -c1-100
find $O -name '*.csv' -delete
