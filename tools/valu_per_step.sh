# VALU / SALU instructions per walk step of the table kernel: tools/valu_per_step.sh SPEC P Q [EF] -> gpurun_out/valu/<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/valu; mkdir -p $O; TAG=${TAG:-run}
python $R/tools/one_walk.py $1 $2 $3 reference 3 ${4:-16} > $O/$TAG.time.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $O/raw -o p --output-format csv -- python $R/tools/one_walk.py $1 $2 $3 reference 2 ${4:-16} > $O/$TAG.pmc.log 2>&1
python - $O/raw $O/$TAG.pmc.log <<'PY' > $O/$TAG.txt
import csv, sys, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_walk_tables' not in k: continue
        acc['k_walk_tables'][r['Counter_Name']] += float(r['Counter_Value'])
steps = sum(int(m) for m in re.findall(r'steps (\d+)', open(sys.argv[2]).read()))
for k, v in acc.items():
    print(k, 'steps', steps)
    for c, x in v.items(): print('  %s %.4g  per step %.1f' % (c, x, x / max(steps, 1)))
PY
rm -rf $O/raw
grep iter $O/$TAG.time.log; cat $O/$TAG.txt
