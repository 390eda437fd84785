R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 3 --graph list --cpu-budget 0 --profile-steps 1 --min-time 0 > $O/trace.log 2>&1
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
tail -25 $O/pytest.log
python3 -c "
import json,sys
d=json.load(open('$O/bench.json'))
print(d['value'],d['ms_per_step'],d['step_modes'])
for r in d['roofline_kernels']: print('  ',r['kernel'][:40],r['launches_per_step'],r['ms_per_step'],r['frac'])
"
tail -5 $O/bench.err
