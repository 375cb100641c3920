cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/$tag -o p --output-format csv -- python $R/tools/one_walk.py 24w 0.25 4 reference 2 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' > $O/$tag.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:60]
    if 'k_walk_tables' not in k and 'k_eb_build' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    print(k)
    for c,x in v.items(): print('  ',c,x)
PY
  rm -rf $O/$tag
done
cat $O/*.txt | grep -v "^$" | head -80
