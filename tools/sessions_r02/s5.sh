set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
python -c "import torch; print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())" > gpurun_out/s5/torch.txt 2>&1; cat gpurun_out/s5/torch.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_mode_a.py -q -m gpu > gpurun_out/s5/pytest.txt 2>&1; tail -25 gpurun_out/s5/pytest.txt
T=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_timing.so
SRW_LIB=$T timeout 900 python tools/explore_edge_tables.py 24w 0.25 4 16 skip > gpurun_out/s5/t24w.txt 2>&1; cat gpurun_out/s5/t24w.txt
timeout 900 python tools/explore_edge_tables.py 24w 0.25 4 16 skip > gpurun_out/s5/24w.txt 2>&1; cat gpurun_out/s5/24w.txt
timeout 1500 python tools/explore_edge_tables.py 26d 4 0.5 27 skip > gpurun_out/s5/c5.txt 2>&1; cat gpurun_out/s5/c5.txt
