#!/bin/bash
# like ab.sh, with extra environment for the experiment side only:  ab2.sh <libname> VAR=VALUE ...
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/ab2_$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in base $name; do
    if [ $v = base ]; then
      env GFV_LIB=$R/gen-fvgn-steady_amd/gfv/libgfv.so timeout 300 python3 $R/bench.py --cpu-budget 0 --min-time 1.5 --graph list --skip-fp32-form --skip-drop-in > $O/${v}_$rep.json 2> $O/${v}_$rep.err
    else
      env GFV_LIB=$R/gen-fvgn-steady_amd/gfv/libgfv_$v.so "$@" timeout 300 python3 $R/bench.py --cpu-budget 0 --min-time 1.5 --graph list --skip-fp32-form --skip-drop-in > $O/${v}_$rep.json 2> $O/${v}_$rep.err
    fi
    python3 -c "
import json
d=json.load(open('$O/${v}_$rep.json'))
print('$v', $rep, d['ms_per_step'], [ (r['kernel'][:14], r['ms_per_step']) for r in d['roofline_kernels'] if 'dw_' in r['kernel'] or 'reduce' in r['kernel']])
"
  done
done
