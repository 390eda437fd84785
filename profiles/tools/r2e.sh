R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2e
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|FAILED|RCCL|Error|assert " $O/pytest.log | head -40
grep -E "^\[|  loss|  uvp|  grad|beyond|worst" $O/pytest.log | head -80
python3 -c "
import json,sys
d=json.load(open('$O/bench.json'))
print(d['value'],d['ms_per_step'],d['step_modes'])
"
