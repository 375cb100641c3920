cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s64; mkdir -p $O
timeout 600 python tools/clock_probe.py 40 26 > $O/clock.txt 2>&1; cat $O/clock.txt | cut -c1-220
rocm-smi --showperflevel --showpowerprofile 2>&1 | grep -v "^=\|^$" | head -12
rocm-smi --showmaxpower --showsclkrange --showmclkrange 2>&1 | grep -v "^=\|^$" | head -12
