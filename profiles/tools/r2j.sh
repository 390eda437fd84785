R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
timeout 300 python3 $R/bench.py --cpu-budget 0 --min-time 1.5 --graph list --skip-fp32-form --skip-drop-in > $O/b_$rep.json 2> $O/b_$rep.err
python3 -c "
import json
d=json.load(open('$O/b_$rep.json'))
print($rep, d['ms_per_step'], d['value'])
"
done
