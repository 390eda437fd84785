#!/bin/bash
# A/B/C... on ONE box: the product library against several experiment builds, alternating, three repetitions
#   gpurun -- 'bash profiles/tools/abn.sh <name1> <name2> ...'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/abn_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in base "$@"; do
    lib=$R/gen-fvgn-steady_amd/gfv/libgfv.so
    [ $v != base ] && lib=$R/gen-fvgn-steady_amd/gfv/libgfv_$v.so
    env GFV_LIB=$lib timeout 300 python3 $R/bench.py --cpu-budget 0 --min-time 1.5 --graph list --skip-fp32-form --skip-drop-in > $O/${v}_$rep.json 2> $O/${v}_$rep.err
    python3 -c "
import json
d=json.load(open('$O/${v}_$rep.json'))
print('$v', $rep, d['ms_per_step'], ' '.join('%s=%.3f' % (r['kernel'].replace('tchain_kernel','tc').replace(', ','')[:18], r['ms_per_step']) for r in d['roofline_kernels'] if r['ms_per_step'] >= 0.02))
"
  done
done
