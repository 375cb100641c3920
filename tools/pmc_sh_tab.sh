# Per-step counters of k_sh_step_tab (config 3's shape on a shard, world 1), variants SRW_SH_BATCH/SRW_SH_GRAB = 0/16 and 2/32:
# separate rocprofv3 --pmc passes (tools/pmc_sh_tab.sh -> gpurun_out/pmc_sh/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sh; mkdir -p $O
run() {  # tag, counters, variant
  tag=$1; ctrs=$2; v=$3
  SKIP_REPLICATED=1 SRW_AB=$v timeout 400 rocprofv3 --kernel-trace --pmc $ctrs -d $O/raw -o p --output-format csv -- python $R/tools/shard_tables_bench.py 24 16 1 0 0.25 4 1 > $O/$tag.log 2>&1
  python - $O/raw $O/$tag.log <<'PY' > $O/$tag.txt
import csv, sys, glob, collections, re
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_sh_step_tab' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value'])
log = open(sys.argv[2]).read()
# the L = 1 walk that builds the tables (2 launches) + the measured walk (81 launches): steps of both
steps = [int(m) for m in re.findall(r'steps (\d+)', log)]
n = steps[0] * 83.0 / 81.0 if steps else 1.0     # (the table-building walk of length 1 adds 2 launches to the 81 of the measured walk)
print('walk-steps of the measured walk', steps[:1])
for c, x in acc.items(): print('%s total %.6g per step (all launches / steps x 83/81) %.2f' % (c, x, x / n))
PY
  rm -rf $O/raw; cat $O/$tag.txt
}
for v in 0/16 2/32; do
  t=$(echo $v | tr / _)
  run sq_$t "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" $v
  run rd_$t TCC_EA0_RDREQ_sum $v
done
