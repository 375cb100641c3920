set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s22; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 300 python tests/fuzz_parity.py 120 > $O/fuzz.txt 2>&1 < /dev/null; tail -3 $O/fuzz.txt
