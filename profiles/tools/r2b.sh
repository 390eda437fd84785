R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for prio in 0 10; do
GFV_SIDE_PRIO=$prio timeout 600 python3 $R/bench.py --cpu-budget 0 > $O/bench_prio$prio.json 2> $O/bench_prio$prio.err
done
tail -5 $O/pytest.log
for prio in 0 10; do python3 -c "
import json,sys
d=json.load(open('$O/bench_prio$prio.json'))
print('prio',$prio,d['value'],d['ms_per_step'],d['step_modes'],d['roofline_step']['priced_share_of_single_stream_step'],d['roofline_step']['frac_compulsory'])
for r in d['roofline_kernels']: print('  ',r['kernel'][:40],r['launches_per_step'],r['ms_per_step'],r['frac'])
"; done
tail -3 $O/bench_prio0.err
