# counters of the table-walk kernel under a given SRW_TABLE_LANES mode: tools/pmc_lanes.sh MODE TAG  -> gpurun_out/pmc_lanes/TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lanes; mkdir -p $O; MODE=$1; TAG=$2
[ -f $O/avail.txt ] || rocprofv3 -L > $O/avail.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  SRW_TABLE_LANES=$MODE timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/raw_$TAG_$i -o p --output-format csv -- python $R/tools/one_walk.py 24w 0.25 4 reference 2 > $O/$TAG.$i.log 2>&1
done
python - $O/$TAG <<'PY' > $O/$TAG.txt
import csv, sys, glob, collections, re
acc = collections.defaultdict(float); steps = 0
for f in glob.glob('/tmp/raw_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_walk_tables' in r['Kernel_Name']: acc[r['Counter_Name']] += float(r['Counter_Value'])
log = open(sys.argv[1] + '.1.log').read()
steps = sum(int(m) for m in re.findall(r'steps (\d+)', log))
print('steps per pass', steps)
for c, x in sorted(acc.items()): print('  %s %.4g  per step %.2f' % (c, x, x / max(steps, 1)))
PY
rm -rf /tmp/raw_*
grep iter $O/$TAG.1.log | head -3; cat $O/$TAG.txt
