set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s27; mkdir -p $O
export SRW_DEBUG_HANDOVER=1
timeout 600 python tools/one_walk.py 24w 0.25 1 reference 2 > $O/q1_24w.txt 2>&1 < /dev/null; grep -E "^iter|handover" $O/q1_24w.txt
timeout 600 python tools/one_walk.py 24 0.25 1 reference 2 > $O/q1_24.txt 2>&1 < /dev/null; grep -E "^iter|handover" $O/q1_24.txt
timeout 600 python tools/one_walk.py 26 0.5 1 reference 2 > $O/q1_26.txt 2>&1 < /dev/null; grep -E "^iter|handover" $O/q1_26.txt
timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/r_24w.txt 2>&1 < /dev/null; grep -E "^iter|handover" $O/r_24w.txt
