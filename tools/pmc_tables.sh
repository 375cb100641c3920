# Per-launch counters of k_walk_tables at config 3 / config 5's stand-in: separate rocprofv3 --pmc passes (tools/pmc_tables.sh -> gpurun_out/pmc2/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; mkdir -p $O
run() {  # tag, counters, one_walk args...
  tag=$1; ctrs=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $O/raw -o p --output-format csv -- python $R/tools/one_walk.py "$@" > $O/$tag.log 2>&1
  python - $O/raw $O/$tag.log <<'PY' > $O/$tag.txt
import csv, sys, glob, collections, re
acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_walk_tables' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Counter_Name'], r['Dispatch_Id'])
        if key not in seen: seen.add(key); n[r['Counter_Name']] += 1
log = open(sys.argv[2]).read()
steps = [int(m) for m in re.findall(r'steps (\d+)', log)]
print('launches', dict(n), 'steps per launch', steps[:1], 'kernel ms', re.findall(r'kernel ([0-9.]+) ms', log))
for c, x in acc.items():
    L = max(n[c], 1); print('%s per launch %.6g per step %.2f' % (c, x / L, x / L / max(steps[0] if steps else 1, 1)))
PY
  rm -rf $O/raw; cat $O/$tag.txt
}
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do run c3_$c $c 24w 0.25 4 reference 2; done
run c3_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" 24w 0.25 4 reference 2
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do run c5_$c $c 26d 4 0.5 reference 1 27; done
run c5_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" 26d 4 0.5 reference 1 27
