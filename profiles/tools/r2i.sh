R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python3 $R/bench.py --cpu-budget 0 --min-time 1.0 --graph list --skip-fp32-form --skip-drop-in > $O/bench_$tag.json 2> $O/bench_$tag.err; python3 -c "
import json
d=json.load(open('$O/bench_$tag.json'))
print('$tag',d['value'],d['ms_per_step'],d['step_modes'])
"; }
run wgs512 GFV_DW_WGS=512
run wgs256 GFV_DW_WGS=256
run wgs128 GFV_DW_WGS=128
run wgs1024 GFV_DW_WGS=1024
run nooverlap GFV_OVERLAP=0
