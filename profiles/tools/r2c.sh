R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 > $O/bench_$tag.json 2> $O/bench_$tag.err; }
run w4 GFV_X=1
run w8big GFV_CHAIN_W_BIG=8
run w8all GFV_CHAIN_W_BIG=8 GFV_CHAIN_W_SMALL=8
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-graph --cpu-budget 0 --profile-steps 1 --min-time 0 > $O/trace.log 2>&1
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
tail -15 $O/pytest.log
for tag in w4 w8big w8all; do python3 -c "
import json,sys
d=json.load(open('$O/bench_$tag.json'))
print('$tag',d['value'],d['ms_per_step'],d['step_modes'])
for r in d['roofline_kernels']: print('  ',r['kernel'][:40],r['launches_per_step'],r['ms_per_step'],r['frac'])
"; done
