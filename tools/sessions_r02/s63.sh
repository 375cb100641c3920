cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s63; mkdir -p $O
for i in 1 2 3; do SRW_TIMING=1 timeout 600 python tools/placement_probe.py 5 26 2>&1 | grep -E "rebuild|bytes at" | tee -a $O/probe.txt; echo "--- new process" | tee -a $O/probe.txt; done
