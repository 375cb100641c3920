set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge_tables or giant_row or binned or karate or rmat_vs" > gpurun_out/s2/pytest_new.txt 2>&1; tail -15 gpurun_out/s2/pytest_new.txt
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/s2/pytest_scale.txt 2>&1; tail -15 gpurun_out/s2/pytest_scale.txt
timeout 300 python tools/explore_edge_tables.py 20 0.25 4 > gpurun_out/s2/ex20.txt 2>&1; cat gpurun_out/s2/ex20.txt
timeout 600 python tools/explore_edge_tables.py 22w 0.25 4 > gpurun_out/s2/ex22w.txt 2>&1; cat gpurun_out/s2/ex22w.txt
timeout 900 python tools/explore_edge_tables.py 24w 0.25 4 16 skip > gpurun_out/s2/ex24w.txt 2>&1; cat gpurun_out/s2/ex24w.txt
