# End-of-round-5 kernel traces (run from the repo root on the GPU box) -> gpurun_out/r05f/: config 3's cold start (table build on
# two streams + side stream) and the embedding stage's training kernel, rocprofv3 --kernel-trace --stats each.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # tag, command...
  tag=$1; shift
  rm -rf $O/raw; timeout 900 rocprofv3 --kernel-trace --stats -d $O/raw -o p --output-format csv -- "$@" > $O/$tag.log 2>&1
  f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $R/tools/prof_summary.py stats $f > $O/${tag}_kernel_stats.txt
  rm -rf $O/raw; head -16 $O/${tag}_kernel_stats.txt
}
stats c3_cold_start python $R/tools/one_walk.py 24w 0.25 4 reference 2
stats embedding python $R/tools/w2v_timing.py 20 1 1
