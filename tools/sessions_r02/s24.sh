set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cli" > $O/pytest_cli.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest_cli.txt; grep -E "^FAILED|Error" $O/pytest_cli.txt | head -5
free -g | head -2; nproc
timeout 1500 python tests/big_c3_check.py 24 > $O/big_c3.txt 2>&1 < /dev/null; grep -v amdgpu $O/big_c3.txt | tail -8
