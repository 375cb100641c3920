cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s78; mkdir -p $O
timeout 900 python -m pytest tests/test_properties.py -q -m gpu > $O/props.txt 2>&1 < /dev/null; tail -3 $O/props.txt | cut -c1-300
