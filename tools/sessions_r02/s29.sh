cd $GRAFT_REPO_ROOT
O=gpurun_out/s29; mkdir -p $O
timeout 300 python tools/debug_links.py > $O/dbg.txt 2>&1 < /dev/null; cat $O/dbg.txt | tail -20
