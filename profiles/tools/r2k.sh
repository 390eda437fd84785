R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_operators_gpu.py tests/test_model_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash profiles/tools/ab.sh oldslice
