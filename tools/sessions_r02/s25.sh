set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s25
mkdir -p $O
summ() { f=$(find "$1" -name "*$2*.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then python $R/tools/prof_summary.py $3 "$f" $4; else echo "no $2 csv under $1"; fi; }
run() { # name, args...
  n=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python $R/tools/one_walk.py "$@" > $O/$n.txt 2>&1 < /dev/null
  echo "== $n: one_walk.py $*" >> $O/summary.txt; grep "^iter" $O/$n.txt >> $O/summary.txt; summ $O/$n kernel_trace stats | head -6 >> $O/summary.txt
}
run c2 20 1 1 reference 4
run q1 24w 0.25 1 reference 3
run c5a 26d 4 0.5 alias 3 27
run c5r 26d 4 0.5 reference 2 27
cat $O/summary.txt
