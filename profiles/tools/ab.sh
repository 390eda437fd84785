#!/bin/bash
# A/B on ONE box (box-to-box variance is ~2 %): alternate the product library and an experiment build (profiles/tools/ab_build.sh)
#   gpurun -- 'bash profiles/tools/ab.sh singlebuf [more env for both...]'
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
O=$R/gpurun_out/ab_$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in base $name; do
    lib=$R/gen-fvgn-steady_amd/gfv/libgfv.so
    [ $v != base ] && lib=$R/gen-fvgn-steady_amd/gfv/libgfv_$v.so
    env GFV_LIB=$lib "$@" timeout 300 python3 $R/bench.py --cpu-budget 0 --min-time 1.5 --graph list --skip-fp32-form --skip-drop-in > $O/${v}_$rep.json 2> $O/${v}_$rep.err
    python3 -c "
import json
d=json.load(open('$O/${v}_$rep.json'))
print('$v', $rep, d['ms_per_step'], [ (r['kernel'][:22], r['ms_per_step']) for r in d['roofline_kernels'][:7]])
"
  done
done
