R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --cpu-budget 0 > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-graph --cpu-budget 0 --profile-steps 1 > $O/trace.log 2>&1
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_trace.csv")))
print(len(rows), rows[0].keys())
PY
tail -3 $O/pytest.log; cat $O/bench.json | head -c 600
