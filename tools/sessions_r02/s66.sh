cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s66; mkdir -p $O
for c in 1 0 1 0; do SRW_CONTIG=$c SRW_TIMING=1 timeout 600 python tools/placement_probe.py 5 26 2>&1 | grep -E "rebuild|bytes at" | sed "s/^/contig=$c /" | tee -a $O/probe.txt; done
