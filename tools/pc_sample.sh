# PC sampling of one_walk.py (rocprofv3 beta): aggregates samples per (code object, offset) -> gpurun_out/pcs/agg_<method>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcs; mkdir -p $O
METHOD=${1:-host_trap}; UNIT=${2:-time}; INTERVAL=${3:-50000}; shift 3
timeout 400 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL --kernel-trace \
  -d $O/raw_$METHOD -o p --output-format csv -- python $R/tools/one_walk.py "$@" > $O/run_$METHOD.log 2>&1
echo "rc=$?" >> $O/run_$METHOD.log
ls -la $O/raw_$METHOD/* >> $O/run_$METHOD.log 2>&1
python - $O/raw_$METHOD $O/agg_$METHOD.txt <<'PY'
import csv, sys, glob, collections, os
d, out = sys.argv[1], sys.argv[2]
with open(out, 'w') as fo:
    for f in glob.glob(d + '/**/*pc_sampling*.csv', recursive=True):
        rd = csv.reader(open(f)); hdr = next(rd)
        fo.write('# ' + f + '\n# ' + ','.join(hdr) + '\n')
        cnt = collections.Counter(); n = 0; keep = []
        idx = {h: i for i, h in enumerate(hdr)}
        for row in rd:
            n += 1
            if n <= 5: fo.write('# ' + ','.join(row) + '\n')
            key = tuple(row[idx[h]] for h in hdr if any(t in h.lower() for t in ('code_object', 'offset', 'inst', 'stall', 'reason', 'issued', 'type')))
            cnt[key] += 1
        fo.write('# rows %d\n' % n)
        for k, c in cnt.most_common(): fo.write('%d\t%s\n' % (c, '\t'.join(k)))
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        rd = csv.DictReader(open(f)); seen = {}
        for r in rd:
            k = r.get('Kernel_Name', '')[:70]; seen[k] = seen.get(k, 0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        for k, v in sorted(seen.items(), key=lambda x: -x[1])[:8]: fo.write('# kernel %s %.1f ms\n' % (k, v / 1e6))
PY
rm -rf $O/raw_$METHOD
head -c 3000 $O/agg_$METHOD.txt; tail -5 $O/run_$METHOD.log
