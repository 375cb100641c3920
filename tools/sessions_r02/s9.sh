set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s9
cat > /tmp/cl.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import _pkg
pkg = _pkg.load()
with pkg.Cluster([0]) as cl:
    cl.generate_rmat(24, 16 << 24, seed=42)
    cl.walk(fetch=False, walk_length=80, num_walks=1, seed=1)
    t = time.time(); st = cl.walk(fetch=False, walk_length=80, num_walks=4, seed=1, batch=4); dt = time.time() - t
    print(f"cluster world 1 batch 4: {st['n_steps']/dt/1e9:.2f} G steps/s ({dt*1e3:.0f} ms)", flush=True)
PY
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s9/prof -o cl -- python /tmp/cl.py > $R/gpurun_out/s9/cl.txt 2>&1
tail -3 $R/gpurun_out/s9/cl.txt
find $R/gpurun_out/s9/prof -name "*kernel_stats*" | head; f=$(find $R/gpurun_out/s9/prof -name "*kernel_stats.csv" | head -1); head -20 $f
