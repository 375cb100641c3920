cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s65; mkdir -p $O
timeout 900 python tools/clock_probe.py 600 26 0 > $O/clock.txt 2>&1; cat $O/clock.txt | cut -c1-200
