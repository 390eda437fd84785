R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2f
mkdir -p $O
cd $R
export GFV_PARITY_REPORT=$O/parity_fp64.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|FAILED|RCCLDIFF|RCCLRESULT" $O/pytest.log | head -40
grep -E "^E  " $O/pytest.log | head -40
python3 -c "
import json,sys
d=json.load(open('$O/bench.json'))
print(d['value'],d['ms_per_step'],d['step_modes'], d['roofline_step']['priced_launch_records_per_step'])
for r in d['roofline_kernels']: print('  ',r['kernel'][:40],r['launches_per_step'],r['ms_per_step'],r['frac'])
"
cat $O/parity_fp64.txt | head -60
