set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
T=$GRAFT_REPO_ROOT/stellar-random-walk_amd/libstellar_rw_timing.so
SRW_LIB=$T timeout 600 python tools/explore_edge_tables.py 22w 0.25 4 > gpurun_out/s3/t22w.txt 2>&1; cat gpurun_out/s3/t22w.txt
SRW_LIB=$T timeout 900 python tools/explore_edge_tables.py 24w 0.25 4 16 skip > gpurun_out/s3/t24w.txt 2>&1; cat gpurun_out/s3/t24w.txt
SRW_EB_MIN_COST=1024 timeout 900 python tools/explore_edge_tables.py 24w 0.25 4 16 skip > gpurun_out/s3/c1024_24w.txt 2>&1; cat gpurun_out/s3/c1024_24w.txt
SRW_EB_MIN_COST=0 timeout 900 python tools/explore_edge_tables.py 24w 0.25 4 16 skip > gpurun_out/s3/c0_24w.txt 2>&1; cat gpurun_out/s3/c0_24w.txt
timeout 1500 python tools/explore_edge_tables.py 26d 4 0.5 27 skip > gpurun_out/s3/c5.txt 2>&1; cat gpurun_out/s3/c5.txt
