R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 > $O/bench.json 2> $O/bench.err
GFV_CSR_FUSE=0 timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 > $O/bench_nofuse.json 2> $O/bench_nofuse.err
grep -E "passed|failed|FAILED" $O/pytest.log | head -40
grep -E "^E  " $O/pytest.log | head -30
for f in bench bench_nofuse; do python3 -c "
import json,sys
d=json.load(open('$O/$f.json'))
print('$f',d['value'],d['ms_per_step'],d['step_modes'], d['roofline_step']['priced_launch_records_per_step'])
for r in d['roofline_kernels']: print('  ',r['kernel'][:40],r['launches_per_step'],r['ms_per_step'],r['frac'])
"; done
tail -3 $O/bench.err
