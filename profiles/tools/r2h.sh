R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_boundary_gpu.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED" $O/pytest.log | head -40
grep -E "^E  " $O/pytest.log | cut -c1-300 | head -60
