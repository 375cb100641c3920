set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/s7/pytest.txt 2>&1; tail -15 gpurun_out/s7/pytest.txt
( time timeout 1500 python bench.py ) > gpurun_out/s7/bench.json 2> gpurun_out/s7/bench.err; tail -5 gpurun_out/s7/bench.err; cat gpurun_out/s7/bench.json | head -c 6000
