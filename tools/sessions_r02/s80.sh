cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s80; mkdir -p $O
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -- python $R/tools/placement_probe.py 6 26 > $O/$n.txt 2>&1 < /dev/null
  python $R/tools/spread_counters.py $O/$n 2>&1 | tee $O/$n.summary.txt | head -40
done
find $O -name '*.csv' -delete
