# Round-6 profile of the bench's kernels (run from the repo root on the GPU box): tools/r06_profile.sh -> gpurun_out/r06prof/
#  1. rocprofv3 --kernel-trace --stats of the headline bench command (the in-run --pmc child passes are switched off inside it: --pmc 0)
#  2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ_sum) per launch of the table kernel of config 3 (k_walk_tables_lanes) and config 5's stand-in (k_walk_tables)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # tag, command...
  tag=$1; shift
  rm -rf $O/raw; timeout 900 rocprofv3 --kernel-trace --stats -d $O/raw -o p --output-format csv -- "$@" > $O/$tag.log 2>&1
  f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $R/tools/prof_summary.py stats $f > $O/${tag}_kernel_stats.txt
  rm -rf $O/raw; head -14 $O/${tag}_kernel_stats.txt
}
pmc() {   # tag, kernel substring, counter, one_walk args...
  tag=$1; kern=$2; ctr=$3; shift 3
  rm -rf $O/raw; timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $O/raw -o p --output-format csv -- python $R/tools/one_walk.py "$@" > $O/${tag}_$ctr.log 2>&1
  python - $O/raw $O/${tag}_$ctr.log $kern <<'PY' > $O/${tag}_$ctr.txt
import csv, sys, glob, collections, re
acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[3] not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Counter_Name'], r['Dispatch_Id'])
        if key not in seen: seen.add(key); n[r['Counter_Name']] += 1
log = open(sys.argv[2]).read()
steps = [int(m) for m in re.findall(r'steps (\d+)', log)]
print('kernel', sys.argv[3], 'launches', dict(n), 'steps per launch', steps[:1], 'kernel ms', re.findall(r'kernel ([0-9.]+) ms', log))
for c, x in acc.items():
    L = max(n[c], 1); print('%s per launch %.6g per step %.3f' % (c, x / L, x / L / max(steps[0] if steps else 1, 1)))
PY
  rm -rf $O/raw; cat $O/${tag}_$ctr.txt
}
stats headline python $R/bench.py --steps 10 --warmup 2 --configs 0 --cpu-baseline 0 --end-to-end 0 --pmc 0 --detail $O/headline_bench_detail.json
grep -o '^{.*' $O/headline.log | tail -1 > $O/headline_bench_line.json
stats c3_cold_start python $R/tools/one_walk.py 24w 0.25 4 reference 2
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do pmc c3 k_walk_tables $c 24w 0.25 4 reference 2; done
for c in FETCH_SIZE TCC_EA0_RDREQ_sum; do pmc c5 k_walk_tables $c 26d 4 0.5 reference 1 27; done
git -C $R rev-parse --short HEAD 2>/dev/null > $O/commit.txt
